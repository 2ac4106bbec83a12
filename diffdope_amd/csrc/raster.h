// raster.h -- scratch layout and entry points of the rasteriser (raster.hip), shared with the renderer
// ops and the fused engine.
#pragma once
#include "raster_math.h"

#define RASTER_SMALL_PX 64    // bbox of at most this many pixel centres => resolved in scatter_kernel (16 / 32 / 64: midpoly 4.7k / 7.5k / 7.6k it/s, lowpoly 10.7k / 9.8k / 11.2k, cfg2 unchanged)
#define RASTER_BIG_GRID 1024  // workgroups of the large-triangle pass

struct RasterScratch {
    int* counters;            // [16]: 3 = large triangles of this pass
    int* tile_flag;           // [B,NT] != 0: tile holds or borders a possibly covered pixel (plain stores of 1)
    int* tile_big;            // [B,NT] != 0: a large triangle overlaps the tile
    int* active;              // [B,NT] per-hypothesis ordered list of active tiles, packed ty << 16 | tx (first b_count[b] entries)
    int* b_count;             // [B] active tiles of each hypothesis
    int2* snap;               // [B,V] window coordinates in 1/256 px (x = INT_MIN if w <= 0)
    uint2* biglist;           // [B,T] the LARGE triangles of each hypothesis: (triangle id, packed tile range tx0 | ty0<<8 | (nx-1)<<16 | (ny-1)<<24)
    int* bigcount;            // [B] entries of biglist (appended by scatter_kernel, one atomic per wave; re-armed by the consumer)
    unsigned long long* zbuf; // [B, zper] (depth key << 32 | triangle id), all ones = background; per hypothesis the frame is
                              // stored in 4x4-pixel blocks (16 entries = one 128-byte line), see zaddr()
    size_t zbuf_bytes;
    size_t zper;              // entries per hypothesis = zwb * ceil(H/4) * 16
    int zwb;                  // 4x4 blocks per row = ceil(W/4)
    const int4* trisort;      // [T] or null: the triangles in the processing order of scatter_kernel, {v0, v1, v2, original id}
    // Back-face culling for CLOSED meshes (fused engine only; the op-level entry draws both faces like nvdiffrast).  A closed,
    // consistently oriented surface that lies entirely inside the view volume covers every pixel centre with as many front- as
    // back-facing triangles, and the nearest one is front-facing: skipping the back faces changes nothing in exact arithmetic
    // (deviation D5) and halves the fragments.  cull_sign: 0 = off; +1 / -1 = triangles whose SNAPPED area
    // has this sign are back faces (sign(signed volume) * sign(det of proj's x,y,w rows), decided once per engine on the host).
    // cull_ok [B,8] (null = never): per vertex slice of the transform, 1 when every vertex it produced has w > 0 and
    // -w <= z <= w; a hypothesis culls only while all its slices say so (else its drawn surface may be open).
    int cull_sign;
    const int* cull_ok;
    int scatter_exchange;     // 1: expect more than ~1 covered centre per triangle -- scatter_kernel's fragment-exchange variant
                              // (lane j takes record j; ids in zbuf stay the original ones, so the result does not depend on it);
                              // 2: the compacting variant (long launches in the micro-polygon regime, see scatter_kernel)
    size_t zero_bytes;        // bytes from `counters` that must be zero before a pass (counters + tile_flag + tile_big + bigcount)
    int ntx, nty, NT;
    PixNdc ndc;               // pixel index -> NDC centre constants for (H, W)
#ifdef DDX_TRACE
    unsigned long long* trace;
#endif
};

#ifdef DDX_TRACE
#define DDX_TRACE_BEGIN() const unsigned long long ddx_t0 = __builtin_amdgcn_s_memrealtime()
#define DDX_TRACE_END(buf, kidx, info)                                                                                   \
    do {                                                                                                                 \
        if (threadIdx.x == 0 && (buf)) {                                                                                 \
            const size_t wg = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);                    \
            if (wg < 8192) {                                                                                             \
                unsigned long long* q = (buf) + ((size_t)(kidx) * 8192 + wg) * 4;                                        \
                q[0] = ddx_t0;                                                                                           \
                q[1] = __builtin_amdgcn_s_memrealtime();                                                                 \
                q[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |                 \
                       ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);         \
                q[3] = (info);                                                                                           \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#else
#define DDX_TRACE_BEGIN()
#define DDX_TRACE_END(buf, kidx, info)
#endif

size_t raster_layout(RasterScratch& L, void* base, int B, int V, int T, int H, int W);

// zbuf address of pixel (px, py) inside one hypothesis' frame.  4x4 blocks instead of rows: the 64-bit atomicMin stream of
// the rasteriser is bound by the number of distinct 128-byte lines an instruction touches (tools/ubench/atomic_density.hip:
// 48 ps per lane-atomic on random lines, 9 ps on one line), and the fragments of a small triangle and of its neighbours in
// the mesh form a compact 2-D patch -- 16x1 row segments cut it into twice as many lines as 4x4 blocks do.
__device__ __forceinline__ unsigned zaddr(int px, int py, int zwb)
{
    return (unsigned)(((__mul24(py >> 2, zwb) + (px >> 2)) << 4) | ((py & 3) << 2) | (px & 3));
}
// window-coordinate snap of clip positions (the fused engine does this inside its transform kernel)
int raster_snap(const float* pos, int B, int V, int H, int W, const RasterScratch& L, hipStream_t s);
// scatter + (compaction | large-triangle raster) (no emit); asynchronous on s.  Needs L.snap filled.
// clear: memset counters/tile_flag and re-arm zbuf first (the engine maintains both itself).
// ev (nullable): 2 events recorded before the scatter / compact_big launches (profiling)
int raster_run(const float* pos, const int* tri, int B, int V, int T, int H, int W, const RasterScratch& L, hipStream_t s,
               bool clear, hipEvent_t* ev);

// Active-tile bookkeeping without atomics on hot words (device-scope atomics on one address serialise
// across the 8 XCDs: a per-tile counter cost 27 us per pass here): scatter_kernel flags tiles with plain
// stores; compact_kernel (one workgroup per hypothesis) ballot-compacts each hypothesis' flags into its own
// ordered segment active[b*NT ...] and writes the count b_count[b]; consumers rebuild the global
// enumeration from the B counts with a block scan in LDS (work_prefix) and map work item -> (b, tile)
// with a binary search (work_lookup).
#define WORK_MAX_B 4096

// prefix[0..B] (exclusive scan of b_count) into LDS; returns the total.  All 256 threads must call.
__device__ __forceinline__ int work_prefix(const int* __restrict__ b_count, int B, int* prefix, int* wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int start = 0; start < B; start += 256) {
        const int i = start + tid;
        const int c = i < B ? b_count[i] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (i < B) prefix[i] = carry + woff + incl - c;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (tid == 0) prefix[B] = carry;
    __syncthreads();
    return carry;
}

// largest b with prefix[b] <= w  (w < prefix[B])
__device__ __forceinline__ int work_lookup(const int* prefix, int B, int w)
{
    int lo = 0, hi = B;  // invariant: prefix[lo] <= w < prefix[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= w) lo = mid; else hi = mid;
    }
    return lo;
}
